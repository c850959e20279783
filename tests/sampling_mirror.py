"""Host mirrors of the sampling branch of Engine.Generate (go/main.go:177-200, :294-408) for the tests.

go_*      : the Go code's arithmetic literally -- float32 everywhere, exp in float64, every sum one left-to-right
            chain; the descending sort is stable (Go's sort.Slice leaves ties unspecified).
device_*  : the same algorithm with the summation ORDER of nanollama_amd/csrc/nl_sample.h (256-element tree
            partials for the normaliser; cumulative sums = wavefront prefix + lane prefix + in-chunk chain), so the device result can
            be checked for exact equality.
"""
import numpy as np

F = np.float32


def apply_penalty(logits, recent, penalty, vocab):
    lg = logits.copy()
    if penalty > 1.0:
        for tok in recent:                      # once per OCCURRENCE, in window order (go/main.go:178-186)
            if 0 <= tok < vocab:
                lg[tok] = F(lg[tok] / F(penalty)) if lg[tok] > 0 else F(lg[tok] * F(penalty))
    return lg


def push_recent(recent, tok, window):
    r = list(recent) + [tok]
    if len(r) > window:
        r = r[1:]
    return r


def _probs(lg, temp):
    z = ((lg - lg.max()) / F(temp)).astype(np.float32)
    return np.exp(z.astype(np.float64)).astype(np.float32)


def go_top_p(lg, temp, top_p, u):
    if temp <= 0:
        return int(np.argmax(lg))
    p = _probs(lg, temp)
    total = np.cumsum(p, dtype=np.float32)[-1]          # sequential float32 sum
    p = (p * (F(1.0) / total)).astype(np.float32)
    order = np.argsort(-p, kind="stable")
    cs = np.cumsum(p[order], dtype=np.float32)
    hit = np.nonzero(cs >= F(top_p))[0]
    if len(hit) == 0:
        return int(order[0])
    cut = int(hit[0])
    r = F(u) * cs[cut]
    j = np.nonzero(r <= cs[:cut + 1])[0]
    return int(order[j[0]]) if len(j) else int(order[0])


def go_top_k(lg, temp, top_k, u):
    if temp <= 0:
        return int(np.argmax(lg))
    k = min(top_k, lg.size)
    order = np.argsort(-lg, kind="stable")[:k]           # insertion list: strict '>' keeps the earlier index first
    vals = lg[order]
    probs = np.exp(((vals - vals[0]) / F(temp)).astype(np.float32).astype(np.float64)).astype(np.float32)
    cs = np.cumsum(probs, dtype=np.float32)
    r = F(u) * cs[-1]
    j = np.nonzero(r <= cs)[0]
    return int(order[j[0]]) if len(j) else int(order[0])


def _tree256(x):
    n = (x.size + 255) // 256 * 256
    y = np.zeros(n, np.float32)
    y[:x.size] = x
    y = y.reshape(-1, 256)
    while y.shape[1] > 1:
        y = (y[:, 0::2] + y[:, 1::2]).astype(np.float32)
    return y[:, 0]


def _ceil_mul(f, x):
    """exact ceil(f * x) for a float32 f in [0, 1] and a Python int x (samp_ceil_mul, nl_sample.h)"""
    num, den = float(F(f)).as_integer_ratio()
    return -((-num * int(x)) // den)


def device_top_p(lg, temp, top_p, u):
    """nl_sample.h's top-p (samp_select_radix_kernel / _stream_kernel), restated with Python integers: weights
    W_i = floor(p_i * 2^45), order = p descending / id ascending, cut = first j with CDF_j >= ceil(top_p * TOTAL), pick =
    first j with CDF_j >= max(1, ceil(u * CDF_cut)).  Returns (token, boundary_margin): margin = relative distance of the
    two threshold tests from their boundaries -- tiny margins are where the Go float32 chain may legitimately disagree."""
    if temp <= 0:
        return int(np.argmax(lg)), 1.0
    p = _probs(lg, temp)
    order = np.argsort(-p, kind="stable")
    w = [int(x) for x in np.floor(p[order].astype(np.float64) * 2.0 ** 45)]
    cs, acc = [], 0
    for x in w:
        acc += x
        cs.append(acc)
    total = cs[-1]
    xcut = max(1, _ceil_mul(top_p, total))
    cut = next(j for j, c in enumerate(cs) if c >= xcut)
    xr = max(1, _ceil_mul(u, cs[cut]))
    pick = next(j for j, c in enumerate(cs) if c >= xr)
    r = float(F(u)) * cs[cut]
    margin = min(abs(cs[cut] / total - top_p), abs(cs[pick] - r) / total, abs(cs[pick - 1] - r) / total if pick > 0 else 1.0)
    return int(order[pick]), margin


