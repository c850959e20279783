/* abi_driver.c -- a plain C (C99, no ctypes, no C++) caller of libnanollama_hip.so.
 *
 * Proves that include/nanollama_hip.h compiles as C and that the library can be driven the way a cgo shim would
 * drive it (integration/go/hip_backend.go): nl_create -> nl_upload_tensor per tensor -> nl_finalize -> nl_forward.
 * The model and the expected logits come in one flat file written by tests/test_abi_c_driver.py from a committed
 * golden GGUF (the reference-Python logits of tests/golden/): this program parses no GGUF.
 *
 *   file := "NLDRV1\0\0" | nl_config (17 x 4 bytes) | int32 n_tensors | tensor* | int32 n_tokens | int32 tokens[n] |
 *           float logits[n][vocab]
 *   tensor := int32 name_len | name | uint32 type | uint64 rows | uint64 cols | uint64 nbytes | bytes
 *
 *   usage: abi_driver <file> <tolerance>      exit status 0 = every logit within tolerance and argmax equal
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nanollama_hip.h"

static const unsigned char *take(const unsigned char **p, size_t n) {
    const unsigned char *q = *p;
    *p += n;
    return q;
}

int main(int argc, char **argv) {
    FILE *f;
    long size;
    unsigned char *buf;
    const unsigned char *p;
    nl_config cfg;
    nl_handle h = NULL;
    int32_t n_tensors, n_tokens, t, i;
    const int32_t *tokens;
    const float *want;
    float *logits;
    double worst = 0.0, tol;
    int rc, bad = 0;

    if (argc != 3) { fprintf(stderr, "usage: %s <file> <tolerance>\n", argv[0]); return 2; }
    tol = atof(argv[2]);
    f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    size = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = (unsigned char *)malloc((size_t)size);
    if (!buf || fread(buf, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    p = buf;
    if (memcmp(take(&p, 8), "NLDRV1\0\0", 8) != 0) { fprintf(stderr, "bad magic\n"); return 2; }
    if (sizeof(nl_config) != 17 * 4) { fprintf(stderr, "nl_config is %u bytes, expected 68\n", (unsigned)sizeof(nl_config)); return 2; }
    memcpy(&cfg, take(&p, sizeof cfg), sizeof cfg);
    printf("abi %d, %d device(s), build %s\n", nl_abi_version(), nl_device_count(), nl_build_info());
    if ((rc = nl_create(&cfg, &h)) != NL_OK) { fprintf(stderr, "nl_create: %d %s\n", rc, nl_last_error(NULL)); return 1; }
    memcpy(&n_tensors, take(&p, 4), 4);
    for (t = 0; t < n_tensors; t++) {
        int32_t name_len;
        char name[256];
        uint32_t type;
        uint64_t rows, cols, nbytes;
        const void *data;
        memcpy(&name_len, take(&p, 4), 4);
        if (name_len <= 0 || name_len >= (int32_t)sizeof name) { fprintf(stderr, "bad tensor name\n"); return 2; }
        memcpy(name, take(&p, (size_t)name_len), (size_t)name_len);
        name[name_len] = 0;
        memcpy(&type, take(&p, 4), 4);
        memcpy(&rows, take(&p, 8), 8);
        memcpy(&cols, take(&p, 8), 8);
        memcpy(&nbytes, take(&p, 8), 8);
        data = take(&p, (size_t)nbytes);
        if ((rc = nl_upload_tensor(h, name, type, data, nbytes, rows, cols)) != NL_OK) {
            fprintf(stderr, "nl_upload_tensor %s: %d %s\n", name, rc, nl_last_error(h));
            return 1;
        }
    }
    if ((rc = nl_finalize(h)) != NL_OK) { fprintf(stderr, "nl_finalize: %d %s\n", rc, nl_last_error(h)); return 1; }
    if ((rc = nl_get_config(h, &cfg)) != NL_OK) return 1;
    memcpy(&n_tokens, take(&p, 4), 4);
    tokens = (const int32_t *)take(&p, (size_t)n_tokens * 4);
    want = (const float *)p;
    logits = (float *)malloc((size_t)cfg.vocab * sizeof(float));
    if ((rc = nl_reset(h, 0)) != NL_OK) return 1;
    for (t = 0; t < n_tokens; t++) {
        int best_got = 0, best_want = 0, fast = -1;
        if ((rc = nl_forward(h, 0, tokens[t], t, logits)) != NL_OK) { fprintf(stderr, "nl_forward: %d %s\n", rc, nl_last_error(h)); return 1; }
        for (i = 0; i < cfg.vocab; i++) {
            double d = fabs((double)logits[i] - (double)want[(size_t)t * cfg.vocab + i]);
            if (d > worst) worst = d;
            if (logits[i] > logits[best_got]) best_got = i;
            if (want[(size_t)t * cfg.vocab + i] > want[(size_t)t * cfg.vocab + best_want]) best_want = i;
        }
        /* the greedy fast path (no logits across the bus) must name the same token */
        if ((rc = nl_forward_argmax(h, 0, tokens[t], t, &fast)) != NL_OK) return 1;
        if (best_got != best_want || fast != best_got) { fprintf(stderr, "pos %d: argmax %d / %d / fast %d\n", t, best_got, best_want, fast); bad = 1; }
    }
    /* errors are values, never crashes: a token outside the vocabulary is refused */
    if (nl_forward(h, 0, cfg.vocab, 0, logits) != NL_ERR_INVALID) { fprintf(stderr, "out-of-range token accepted\n"); bad = 1; }
    printf("max |logit - golden| = %.3g over %d positions (tolerance %.3g)\n", worst, n_tokens, tol);
    nl_destroy(h);
    free(logits);
    free(buf);
    return (bad || !(worst <= tol)) ? 1 : 0;
}
