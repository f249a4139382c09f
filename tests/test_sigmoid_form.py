"""The output sigmoid of the collision branch's MLP (csrc/tgnn_common.h: sigmoid_out_f32; reference: torch.sigmoid behind the last
Linear of GINConv's MLP, /root/reference/graph_networks/layers/coll_conv.py:14-18) restated in numpy float32 with the constants
READ FROM THE HEADER: the two-part product -v log2 e = th + tl, e = 2^th (1 + tl ln 2), r = 1 / (1 + e) with one Newton step.
On the host exp2 and the reciprocal seed are correctly rounded where the device's v_exp_f32 / v_rcp_f32 are 1-ulp approximations,
so this pins the FORM (the split of log2 e, the correction term, the Newton step, the clamp), not the hardware: against fp64 the
form itself must be as close as a correctly rounded fp32 exp + an IEEE division is (both ~3 ulp at worst: 1 ulp of relative error
in e is up to 2 ulp of the quotient), and the one-part product it replaces must be visibly worse.  The device's result is held to the fp64 oracle by
tests/test_gin_fused.py (GINConv rows, 1e-6 of the largest value; measured 1.2e-7 .. 1.4e-7)."""
import os
import re

import numpy as np

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tilingnn_amd", "csrc", "tgnn_common.h")


def header_constants():
    src = open(HEADER).read()
    body = src[src.index("float sigmoid_out_f32(float v)"):]
    body = body[:body.index("\n}\n")]
    vals = {k: np.float32(float(v)) for k, v in re.findall(r"(kL2eH|kL2eL|kLn2) = ([0-9.eE+-]+)f", body)}
    assert set(vals) == {"kL2eH", "kL2eL", "kLn2"}, vals
    clamp = re.search(r"fmaxf\(v, (-[0-9.]+)f\)", body)
    assert clamp, "the clamp of the argument"
    return vals, np.float32(float(clamp.group(1)))


def fma32(a, b, c):
    """fmaf on float32 arrays: the product and the sum in fp64 (exact for fp32 operands), one rounding"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def sigmoid_form(v, k, clamp):
    one = np.float32(1.0)
    nv = -np.minimum(np.maximum(v, clamp), -clamp)          # (both sides: e stays finite and non-zero)
    th = nv * k["kL2eH"]
    tl = fma32(nv, np.full_like(nv, k["kL2eH"]), -th) + nv * k["kL2eL"]
    eh = np.exp2(th.astype(np.float64)).astype(np.float32)
    e = fma32(eh, tl * k["kLn2"], eh)
    d = one + e
    r = (1.0 / d.astype(np.float64)).astype(np.float32)
    r = fma32(fma32(-d, r, np.full_like(d, one)), r, r)
    return np.where(np.isnan(v), v, r)                      # (the clamps drop a NaN: put back, as torch.sigmoid propagates it)


def test_constants_are_log2e_in_two_parts():
    k, clamp = header_constants()
    log2e = 1.0 / np.log(2.0)
    assert k["kL2eH"] == np.float32(log2e)
    assert abs(float(k["kL2eH"]) + float(k["kL2eL"]) - log2e) < 1e-15            # hi + lo carries 48 bits of log2 e
    assert k["kLn2"] == np.float32(np.log(2.0))
    assert -88.0 < float(clamp) <= -80.0                                          # exp(-clamp) finite in fp32


def test_form_is_as_close_to_fp64_as_exp_and_division():
    k, clamp = header_constants()
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.uniform(-30, 30, 400_000), rng.normal(0, 3, 400_000), np.linspace(-100, 100, 20_001),
                        [0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0, 1e4, -1e4]]).astype(np.float32)
    got = sigmoid_form(v, k, clamp)
    with np.errstate(over="ignore"):                                            # (exp(100) in fp64 is fine, exp(1e4) is inf: sigmoid 0)
        want = 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
    normal = want > 1e-37
    ulp = np.spacing(want[normal].astype(np.float32)).astype(np.float64)
    err = np.abs(got[normal].astype(np.float64) - want[normal]) / ulp
    with np.errstate(over="ignore"):
        libm = (np.float32(1.0) / (np.float32(1.0) + np.exp(-v[normal]))).astype(np.float32)  # expf + division in fp32
    err_libm = np.abs(libm.astype(np.float64) - want[normal]) / ulp
    print(f"max error in ulp: two-part form {err.max():.2f}, fp32 exp + division {err_libm.max():.2f}")
    assert err.max() <= 3.25 and err.max() <= err_libm.max() + 0.25 and err.mean() < 0.5
    assert (got[~normal] < 2e-37).all()                                           # (the clamped tail: below fp32's normal range)
    # a single-part product (what the two parts are for) loses more than that at |v| ~ 20
    th = (-v * k["kL2eH"]).astype(np.float32)
    with np.errstate(over="ignore"):
        single = (1.0 / (1.0 + np.exp2(th.astype(np.float64)))).astype(np.float32)
    big = normal & (np.abs(v) > 10) & (np.abs(v) < 30)
    err_single = np.abs(single[big].astype(np.float64) - want[big]) / np.spacing(want[big].astype(np.float32))
    print(f"one-part product, 10 < |v| < 30: {err_single.max():.1f} ulp")
    assert err_single.max() > 8.0


def test_monotone_and_symmetric_to_rounding():
    k, clamp = header_constants()
    v = np.linspace(-20, 20, 200_001).astype(np.float32)
    s = sigmoid_form(v, k, clamp)
    assert (np.diff(s.astype(np.float64)) >= -np.spacing(s[1:])).all()
    assert np.abs((s + s[::-1]).astype(np.float64) - 1.0).max() < 2e-7


def test_infinities_and_nan_behave_like_torch_sigmoid():
    """sigmoid(+inf) = 1, sigmoid(-inf) = 0, sigmoid(NaN) = NaN (the one-sided clamp gave NaN at +inf and 1.6e-38 for NaN);
    the header's form carries the two-sided clamp and the NaN pass-through this restatement mirrors."""
    k, clamp = header_constants()
    src = open(HEADER).read()
    body = src[src.index("float sigmoid_out_f32(float v)"):]
    body = body[:body.index("\n}\n")]
    assert re.search(r"fminf\(fmaxf\(v, -[0-9.]+f\), [0-9.]+f\)", body), "two-sided clamp"
    assert "v != v ? v : r" in body, "NaN pass-through"
    with np.errstate(invalid="ignore", over="ignore"):
        got = sigmoid_form(np.array([np.inf, -np.inf, np.nan, 3.0e38, -3.0e38], dtype=np.float32), k, clamp)
    assert got[0] == 1.0 and got[3] == 1.0
    assert 0.0 <= got[1] < 1e-37 and 0.0 <= got[4] < 1e-37
    assert np.isnan(got[2])
