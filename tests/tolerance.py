"""Tolerance reports shared by the GPU parity tests.

`close_report` asserts the worst PURE relative error over every entry with |want| >= floor x max|want| against rtol and an absolute error of
atol_frac x max|want| below that, and prints the numbers so the claim is checkable.  `tiers` is the three-tier bound the single-GPU training
path is held to (tests/test_gpu_fullshape.py): 1e-4 over entries >= 0.1 max, 3e-4 over entries >= 0.01 max, 3e-6 x max below — north_star's
"within 1e-4 relative on float scores" carried through accumulated gradients and updated parameters."""

RTOL = 1e-4
FLOOR = 1e-2


def close_report(got, want, what, rtol=RTOL, floor=FLOOR, atol_frac=None):
    """worst pure-relative error above floor*max, worst absolute error (as a fraction of max) below it."""
    got, want = got.detach().cpu().double().flatten(), want.detach().cpu().double().flatten()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    mx = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs()
    big = want.abs() >= floor * mx
    rel = (err[big] / want.abs()[big]).max().item() if bool(big.any()) else 0.0
    small = (err[~big].max().item() / mx) if bool((~big).any()) else 0.0
    atol_frac = rtol * floor if atol_frac is None else atol_frac
    print("%-28s worst rel (|want| >= %.0e max) %.2e   worst abs/max below %.2e   max %.3e" % (what, floor, rel, small, mx))
    assert rel <= rtol, "%s: worst relative error %.3e > %.1e over entries >= %.0e x max" % (what, rel, rtol, floor)
    assert small <= atol_frac, "%s: worst small-entry error %.3e x max > %.1e x max" % (what, small, atol_frac)
    return rel, small


def tiers(got, want, what, rtol=RTOL, small_frac=None):
    """the three tiers at rtol (default north_star's 1e-4): PURE relative error <= rtol over entries >= 0.1 max, <= 3 rtol over entries >= 0.01 max,
    absolute error <= 0.03 rtol x max (3e-6 x max at 1e-4) below — no `atol = rtol x max` term over the entries that carry the result.
    small_frac: another bound for the entries below 1 % of the maximum, for quantities whose SMALL values are ill-conditioned in the reference's own
    formula (the L2 comparator's sqrt(x^2 + y^2 - 2 x y), comparators.cpp:30-39: a cancellation the reference's fp32 evaluation suffers alike)"""
    close_report(got, want, what, rtol=rtol, floor=0.1, atol_frac=1.0)
    close_report(got, want, what, rtol=3 * rtol, floor=0.01, atol_frac=0.03 * rtol if small_frac is None else small_frac)


# Trajectories of several Adagrad steps from an all-zero state (tests of whole epochs): Adagrad divides every gradient coordinate by its OWN
# history, w -= lr g_k / sqrt(sum g_k^2), so the RELATIVE error of a coordinate — 1e-7 x max|g| / |g_k|: 1e-4 for a coordinate 1e-3 of the
# largest, in the reference's fp32 evaluation as in ours — arrives in the weight as an absolute error of lr times that.  Such tests therefore hold
# the tiers at TRAJECTORY_RTOL and leave out the coordinates whose accumulated g^2 lies more than three decades below the typical one.
TRAJECTORY_RTOL = 1e-3


def well_conditioned(state_ref, floor=1e-8, rel=None):
    """Mask of the table elements whose trajectory is a well-conditioned function of the gradients.  Adagrad from an all-zero sum moves a weight
    by lr g / (|g| + 1e-10) (batch.cpp:67-69): where a first gradient is rounding noise around zero — exactly 0 in one correct fp32 evaluation,
    7e-13 in another — the two updates differ by up to lr (7e-13 -> 7e-4 of a step of 0.1), in ANY arithmetic.  Elements of touched rows whose
    accumulated g^2 stayed below `floor` are such noise and are left out; rows nobody touched (state 0 throughout) stay in and must match exactly.
    rel: the floor is at least rel x the median accumulated g^2 of the touched elements (tables whose gradients are small throughout)."""
    touched_rows = (state_ref > 0).any(1, keepdim=True)
    if rel is not None and bool((state_ref > 0).any()):
        floor = max(floor, rel * float(state_ref[state_ref > 0].median()))
    return (~touched_rows).expand_as(state_ref) | (state_ref > floor)


# |difference of a gradient coordinate| / max |gradient| between two float32-class evaluations of one batch: each is 1-2e-6 from float64 (bench.py
# arith_check: `worst_abs_max`), two of them differ by up to twice that, and a factor 2 of margin
GRAD_NOISE = 8e-6


def first_touch(first, state):
    """running record of every element's accumulated g^2 right after the step that first made it non-zero (call after each reference step)"""
    import torch

    return torch.where((first == 0) & (state > 0), state, first)


def trajectory_close(got, want, state_ref, lr, steps, what, rtol=TRAJECTORY_RTOL, first_state=None):
    """Weights after `steps` Adagrad steps from an all-zero state, element by element, with the bound the update rule itself implies instead of a mask:
        |got - want| <= rtol |want| + 0.03 rtol max|want|                  (the tiers' relative and small-entry terms)
                        + min(lr steps, lr steps GRAD_NOISE sqrt(max S / S_i))   (conditioning of w -= lr g / sqrt(S): batch.cpp:67-69)
    where S = the reference path's accumulated g^2.  A coordinate whose gradients were as large as any gets 3e-6 of slack at lr 0.1, eight steps;
    one whose gradients were 1e-3 of the largest gets 3e-3; one whose S is rounding noise gets the whole lr steps (its first step is lr sign(noise)
    in any arithmetic).  Rows nobody touched (S = 0 throughout the row) must be EQUAL.
    first_state (first_touch): S right after an element's FIRST non-zero gradient.  Step k divides by sqrt(S_k), the sum SO FAR, so an element whose
    first gradient was small and whose later ones were large is as ill-conditioned as its first step, whatever the final sum says: with it the
    conditioning term uses S_first instead of the final S."""
    import torch

    got, want, S = got.detach().cpu().double(), want.detach().cpu().double(), state_ref.detach().cpu().double()
    assert got.shape == want.shape == S.shape, (what, got.shape, want.shape, S.shape)
    if S.dim() == 1:
        got, want, S = got[None], want[None], S[None]
    touched = (S > 0).any(1, keepdim=True).expand_as(S)
    assert torch.equal(got[~touched], want[~touched]), "%s: a row nobody touched changed" % what
    smax, mx = float(S.max()), max(float(want.abs().max()), 1e-30)
    cond = torch.full_like(S, float(lr * steps))
    pos = S > 0
    Sc = S
    if first_state is not None:
        Sc = first_state.detach().cpu().double().reshape(S.shape)
    cond[pos] = torch.clamp(lr * steps * GRAD_NOISE * torch.sqrt(smax / Sc[pos].clamp_min(1e-300)), max=lr * steps)
    allowed = rtol * want.abs() + 0.03 * rtol * mx + cond
    err = (got - want).abs()
    worst = float((err / allowed)[touched].max()) if bool(touched.any()) else 0.0
    tight = float((err <= rtol * want.abs() + 0.03 * rtol * mx)[touched].double().mean()) if bool(touched.any()) else 1.0
    print("%-28s worst error / bound %.3f   (%.1f %% of the touched elements inside the relative terms alone)   max %.3e" % (what, worst, 100 * tight, mx))
    assert worst <= 1.0, "%s: an element is %.2f x outside its conditioning bound" % (what, worst)
    assert tight > 0.9, "%s: only %.1f %% of the touched elements inside the relative terms" % (what, 100 * tight)
