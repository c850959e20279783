"""First-principles known-answer vectors for the sampling branch of Engine.Generate (go/main.go:177-200, :294-398).

Eight logits whose softmax is easy to do by hand; every expected pick sits >= 9e-3 away from the nearest
cumulative-probability boundary, so float32 / float64 rounding and summation order cannot move it.  The numbers in
the comments are the hand calculation; `expected_*` below recompute them in plain Python floats (math.exp, one
left-to-right chain -- the Go loop written out, no numpy, none of the mirrors) and the tests assert that this
recomputation, the literal-Go mirror (sampling_mirror.go_*) and the DEVICE all give the hard-coded answers.
"""
import math

LOGITS = [2.0, 1.0, 0.0, -1.0, 0.5, -3.0, 1.5, -0.5]

# temp 1.0: exp(l - 2) = 1, .36788, .13534, .04979, .22313, .00674, .60653, .08208; sum 2.47149
#   normalised           .40462 .14885 .05476 .02014 .09028 .00273 .24541 .03321
#   descending: id 0 (.40462), 6 (.24541), 1 (.14885), 4 (.09028), 2 (.05476), 7, 3, 5
#   cumulative: .40462 .65003 .79888 .88916 .94392 >= 0.9 -> five candidates, r = u * .94392
TOP_P = [  # (temp, top_p, u, expected id)
    (1.0, 0.9, 0.10, 0),    # r = .09439 <= .40462
    (1.0, 0.9, 0.50, 6),    # r = .47196 in (.40462, .65003]
    (1.0, 0.9, 0.80, 1),    # r = .75514 in (.65003, .79888]
    (1.0, 0.9, 0.90, 4),    # r = .84953 in (.79888, .88916]
    (1.0, 0.9, 0.99, 2),    # r = .93448 in (.88916, .94392]
    # temp 0.5: exp(2(l - 2)) = 1, .13534, .01832, .00248, .04979, .00005, .36788, .00674; sum 1.58059
    #   p(id 0) = .63268 >= 0.6 -> one candidate whatever u is
    (0.5, 0.6, 0.20, 0), (0.5, 0.6, 0.70, 0), (0.5, 0.6, 0.97, 0),
    (0.0, 0.9, 0.50, 0),    # temp 0 routes through argmax (go/main.go:350-352)
]

# top-k 3, temp 0.5: insertion list = ids 0 (2.0), 6 (1.5), 1 (1.0); exp((l - 2) / .5) = 1, .36788, .13534; sum 1.50322
TOP_K = [  # (temp, k, u, expected id)
    (0.5, 3, 0.50, 0),      # r = .75161 <= 1
    (0.5, 3, 0.70, 6),      # r = 1.05225 in (1, 1.36788]
    (0.5, 3, 0.95, 1),      # r = 1.42806 in (1.36788, 1.50322]
    (0.5, 1, 0.99, 0),      # k = 1: the maximum
]

# ties: strict '>' keeps the EARLIER index ahead in sampleTopK's insertion list (go/main.go:318-325) and argmax
# returns the lowest index (go/main.go:400-408)
TIE_LOGITS = [0.5, 3.0, -1.0, 3.0, 0.0, 3.0, 1.0, -2.0]
TIE_TOP_K = [(1.0, 2, 0.25, 1), (1.0, 2, 0.75, 3), (1.0, 3, 0.90, 5), (0.0, 3, 0.5, 1)]

# repetition penalty 2.0 over the window [0, 0, 5, 3, 4]: once per OCCURRENCE, positive logits divided, others multiplied
PENALTY = 2.0
RECENT = [0, 0, 5, 3, 4]
LOGITS_AFTER_PENALTY = [0.5, 1.0, 0.0, -2.0, 0.25, -6.0, 1.5, -0.5]   # id 0: 2 / 2 / 2; id 5: -3 * 2; id 3: -1 * 2; id 4: .5 / 2


def expected_top_p(logits, temp, top_p, u):
    if temp <= 0:
        return max(range(len(logits)), key=lambda i: (logits[i], -i))
    m = max(logits)
    p = [math.exp((x - m) / temp) for x in logits]
    s = 0.0
    for x in p:
        s += x
    p = [x / s for x in p]
    order = sorted(range(len(p)), key=lambda i: (-p[i], i))
    cum = 0.0
    for n, i in enumerate(order):
        cum += p[i]
        if cum >= top_p:
            r, cdf = u * cum, 0.0
            for j in order[:n + 1]:
                cdf += p[j]
                if r <= cdf:
                    return j
            return order[0]
    return order[0]


def expected_top_k(logits, temp, k, u):
    if temp <= 0:
        return max(range(len(logits)), key=lambda i: (logits[i], -i))
    order = sorted(range(len(logits)), key=lambda i: (-logits[i], i))[:min(k, len(logits))]
    pr = [math.exp((logits[i] - logits[order[0]]) / temp) for i in order]
    s = 0.0
    for x in pr:
        s += x
    r, cdf = u * s, 0.0
    for i, x in zip(order, pr):
        cdf += x
        if r <= cdf:
            return i
    return order[0]
