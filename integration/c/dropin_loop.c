/* dropin_loop.c -- the decode loop an UNPATCHED Go host runs over the cgo shim, restated in C99 for timing.
 *
 * integration/go/model_hip.patch routes (*LlamaModel).Forward to nl_forward: one call per token, the V logits copied into
 * State.Logits, the argmax done by host code (go/main.go:213, :400-408).  bench.py's headline number is nl_decode_greedy
 * (steps chained on the device), which only a host that was changed to call it sees; this file gives the rate of the
 * drop-in as it is -- and of the 4-bytes-per-token variant nl_forward_argmax -- with the loop in compiled code, not in
 * Python.  Built by bench.py (gcc -std=c99 -shared) next to the library; takes a finalized handle.
 */
#include <time.h>

#include "nanollama_hip.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* argmax go/main.go:400-408: strict '>' => lowest index wins ties.  The running maximum is kept in a register: written as
 * `v[i] > v[best]`, gcc -O2 turns the update into a conditional move and every iteration's load of v[best] waits for it (a
 * 6-cycle loop-carried chain, ~95 us for 32000 logits); Go's compiler emits a (well predicted) branch for the same source. */
static int host_argmax(const float *v, int n) {
    int best = 0, i;
    float bv = v[0];
    for (i = 1; i < n; i++)
        if (v[i] > bv) { bv = v[i]; best = i; }
    return best;
}

/* Forward(token, pos) -> logits on the host -> host argmax -> next token; n steps.  Returns an nl_status. */
NL_API int nl_dropin_forward_loop(nl_handle h, int stream, int token, int pos, int n, float *logits, int vocab, int *ids_out,
                                  double *seconds) {
    int i, rc;
    const double t0 = now_s();
    for (i = 0; i < n; i++) {
        if ((rc = nl_forward(h, stream, token, pos + i, logits)) != NL_OK) return rc;
        token = host_argmax(logits, vocab);
        ids_out[i] = token;
    }
    *seconds = now_s() - t0;
    return NL_OK;
}

/* the same loop as the cgo shim runs it: State.Logits IS the library's pinned buffer (nl_host_logits), nl_forward copies
 * nothing; *call_seconds = the time inside nl_forward alone (the rest of a step is the host's argmax) */
NL_API int nl_dropin_forward_inplace_loop(nl_handle h, int stream, int token, int pos, int n, int vocab, int *ids_out, double *seconds,
                                          double *call_seconds) {
    int i, rc;
    float *logits = nl_host_logits(h);
    double in_calls = 0.0;
    const double t0 = now_s();
    if (!logits) return NL_ERR_STATE;
    for (i = 0; i < n; i++) {
        const double c0 = now_s();
        if ((rc = nl_forward(h, stream, token, pos + i, logits)) != NL_OK) return rc;
        in_calls += now_s() - c0;
        token = host_argmax(logits, vocab);
        ids_out[i] = token;
    }
    *seconds = now_s() - t0;
    *call_seconds = in_calls;
    return NL_OK;
}

/* the greedy fast path of the ABI: Forward + argmax on the device, one id back per token */
NL_API int nl_dropin_forward_argmax_loop(nl_handle h, int stream, int token, int pos, int n, int *ids_out, double *seconds) {
    int i, rc;
    const double t0 = now_s();
    for (i = 0; i < n; i++) {
        if ((rc = nl_forward_argmax(h, stream, token, pos + i, &token)) != NL_OK) return rc;
        ids_out[i] = token;
    }
    *seconds = now_s() - t0;
    return NL_OK;
}
